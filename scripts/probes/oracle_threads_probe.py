import sys, time, torch
sys.path.insert(0, ".")
from oracle import vl2_oracle as O
cfg = O.config_videollama2_7b(16); cfg["llm"]["num_hidden_layers"] = 2
keep = lambda n: n.startswith(("model.layers.", "model.norm", "model.embed_tokens", "lm_head"))
sd = O.seeded_state_dict(cfg, 31, only=keep)
emb = (0.5 * torch.randn(1621, 4096)).bfloat16().float()
import os
print("cpus", os.cpu_count())
for n in (8, 16, 24, 32, 48, 64):
    torch.set_num_threads(n)
    with torch.no_grad():
        O.mistral_forward(sd, cfg, emb[:64], 0, None)
        t0 = time.time(); O.mistral_forward(sd, cfg, emb, 0, None); t1 = time.time()
    print(n, "threads:", round(t1 - t0, 2), "s for 2 layers at S=1621", flush=True)
