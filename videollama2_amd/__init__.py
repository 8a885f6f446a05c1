"""videollama2_amd -- MI355X-native implementation of ONE path of DAMO-NLP-SG/VideoLLaMA2: the video-inference hot
path (CLIP-ViT per-frame encoder -> STC connector -> Mistral decoder) behind the reference's own Python seams.
Hand-written gfx950 HIP kernels in libvl2hip.so (csrc/), bound through a C ABI (include/vl2hip.h); PyTorch-ROCm
tensors are storage only.  There is no CPU / eager fallback: a missing library raises."""
from .config import videollama2_7b  # noqa: F401
from .constants import *  # noqa: F401,F403
