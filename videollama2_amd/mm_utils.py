"""Host-side pre/post-processing with the reference's API surface (videollama2/mm_utils.py), re-implemented:
    frame_sample            mm_utils.py:106-129
    expand2square           mm_utils.py:27-38
    process_image           mm_utils.py:91-103
    process_video           mm_utils.py:132-202   (same signature, same return: CPU fp32 [T,3,H,W], same ValueError)
    tokenizer_multimodal_token  mm_utils.py:277-302
    KeywordsStoppingCriteria    mm_utils.py:314-345
Video containers need decord / imageio / cv2 exactly as in the reference; they are imported lazily so that the
ndarray / PIL / frame-directory inputs work on boxes without them.  CPU work, outside the GPU timing window."""
import os

import numpy as np
import torch
from PIL import Image

from .constants import MAX_FRAMES, MODAL_INDEX_MAP, NUM_FRAMES, NUM_FRAMES_PER_SECOND, DEFAULT_IMAGE_TOKEN


def frame_sample(duration, mode="uniform", num_frames=None, fps=None):
    """'uniform': index of the centre of each of `num_frames` equal segments of [0, duration-1], round-half-up via
    np.round(x + 1e-6);  'fps': one frame per (fps // NUM_FRAMES_PER_SECOND) frames, starting half a segment in."""
    if mode == "uniform":
        assert num_frames is not None, "Number of frames must be provided for uniform sampling."
        seg = float(duration - 1) / num_frames
        centres = [(seg * i + seg * (i + 1)) / 2 for i in range(num_frames)]      # same float ops as the reference
        return np.round(np.array(centres) + 1e-6).astype(int)
    if mode == "fps":
        assert fps is not None, "FPS must be provided for FPS sampling."
        seg_len = min(fps // NUM_FRAMES_PER_SECOND, duration)
        return np.arange(seg_len // 2, duration, seg_len, dtype=int)
    raise ImportError(f"Unsupported frame sampling mode: {mode}")


def expand2square(pil_img, background_color):
    """Pad the short side symmetrically with `background_color` so the image becomes square."""
    w, h = pil_img.size
    if w == h:
        return pil_img
    side = max(w, h)
    canvas = Image.new(pil_img.mode, (side, side), background_color)
    canvas.paste(pil_img, ((side - w) // 2, (side - h) // 2))
    return canvas


def _mean_colour(processor):
    return tuple(int(x * 255) for x in processor.image_mean)


def process_image(image_path, processor, aspect_ratio="pad"):
    img = Image.fromarray(np.array(Image.open(image_path).convert("RGB")))
    if aspect_ratio == "pad":
        img = expand2square(img, _mean_colour(processor))
    return processor.preprocess([img], return_tensors="pt")["pixel_values"]


def _load_from_path(video_path, s, e, num_frames):
    if s is not None and e is not None:
        s, e = max(s, 0.0), max(e, 0.0)
        if s > e:
            s, e = e, s
        elif s == e:
            e = s + 1
    if os.path.isdir(video_path):
        files = sorted(os.listdir(video_path))
        fps, total, kind = 3, len(files), "dir"
    elif video_path.endswith(".gif"):
        import imageio
        reader = imageio.get_reader(video_path)
        fps, total, kind = 25, len(reader), "gif"
    else:
        from decord import VideoReader
        reader = VideoReader(video_path, num_threads=2)
        fps, total, kind = reader.get_avg_fps(), len(reader), "video"
    first = 0 if s is None else max(int(s * fps) - 1, 0)
    last = total - 1 if e is None else min(int(e * fps) - 1, total - 1)
    span = list(range(first, last + 1))
    picks = frame_sample(len(span), mode="fps", fps=fps) if num_frames is None else \
        frame_sample(len(span), mode="uniform", num_frames=num_frames)
    chosen = [span[i] for i in picks]
    if kind == "dir":
        return [Image.open(os.path.join(video_path, files[i])) for i in chosen]
    if kind == "gif":
        import cv2
        keep = set(chosen)
        return [Image.fromarray(cv2.cvtColor(fr, cv2.COLOR_RGBA2RGB)) for i, fr in enumerate(reader) if i in keep]
    return [Image.fromarray(fr) for fr in reader.get_batch(chosen).asnumpy()]


def process_video(video_path, processor, s=None, e=None, aspect_ratio="pad", num_frames=NUM_FRAMES):
    """video (path | dir of frames | ndarray [T,H,W,3] u8 | list of ndarray / paths / PIL) -> fp32 [T,3,H,W] (CPU)."""
    if isinstance(video_path, str):
        frames = _load_from_path(video_path, s, e, num_frames)
    elif isinstance(video_path, np.ndarray):
        frames = [Image.fromarray(f) for f in video_path]
    elif isinstance(video_path, list) and isinstance(video_path[0], np.ndarray):
        frames = [Image.fromarray(f) for f in video_path]
    elif isinstance(video_path, list) and isinstance(video_path[0], str):
        frames = [Image.open(f) for f in video_path]
    elif isinstance(video_path, list) and isinstance(video_path[0], Image.Image):
        frames = list(video_path)
    else:
        raise ValueError(f"Unsupported video path type: {type(video_path)}")
    # short clips are padded with black frames (the reference builds them as (W, H, 3) arrays: PIL .size is (W, H))
    while num_frames is not None and len(frames) < num_frames:
        frames.append(Image.fromarray(np.zeros((*frames[-1].size, 3), dtype=np.uint8)))
    frames = frames[:MAX_FRAMES]
    if aspect_ratio == "pad":
        frames = [expand2square(f, _mean_colour(processor)) for f in frames]
    return processor.preprocess(frames, return_tensors="pt")["pixel_values"]


def process_video_u8(video_path, processor, s=None, e=None, aspect_ratio="pad", num_frames=NUM_FRAMES):
    """uint8 ingest (not in the reference): the same frame loading / padding / resize / crop as
    `process_video`, but the arithmetic tail (x/255, (x-mean)/std) is left to the GPU: returns uint8 [T, S, S, 3] (CPU), a
    quarter of the fp32 bytes, which `HipCLIPVisionTower` / `HipSiglipVisionTower` accept directly (the tower normalises in
    registers while building the patch rows, with the processor's own rescale_factor / image_mean / image_std)."""
    import copy
    raw = copy.copy(processor)
    raw.do_rescale = False
    raw.do_normalize = False
    px = process_video(video_path, raw, s=s, e=e, aspect_ratio=aspect_ratio, num_frames=num_frames)   # [T,3,S,S] values 0..255
    if px.dtype != torch.uint8:
        px = px.round().clamp_(0, 255).to(torch.uint8)
    return px.permute(0, 2, 3, 1).contiguous()


def tokenizer_multimodal_token(prompt, tokenizer, multimodal_token=DEFAULT_IMAGE_TOKEN, return_tensors=None):
    """Tokenize the text around each `<video>`/`<image>` tag without special tokens and put the (negative) sentinel id
    of MODAL_INDEX_MAP where the tag stood."""
    sentinel = MODAL_INDEX_MAP.get(multimodal_token, None)
    if sentinel is None:
        ids = tokenizer(prompt, add_special_tokens=False).input_ids
    else:
        ids = []
        for i, chunk in enumerate(prompt.split(multimodal_token)):
            if i > 0:
                ids.append(sentinel)
            ids.extend(tokenizer(chunk, add_special_tokens=False).input_ids)
    if return_tensors is None:
        return ids
    if return_tensors == "pt":
        return torch.tensor(ids, dtype=torch.long)
    raise ValueError(f"Unsupported tensor type: {return_tensors}")


def get_model_name_from_path(model_path):
    parts = model_path.strip("/").split("/")
    return parts[-2] + "_" + parts[-1] if parts[-1].startswith("checkpoint-") else parts[-1]


class KeywordsStoppingCriteria:
    """Stop when the generated ids end with a keyword's ids, or the decoded tail contains a keyword string.
    Callable as criteria(output_ids [B, n], scores) -> bool (all rows), like the reference's StoppingCriteria."""

    def __init__(self, keywords, tokenizer, input_ids):
        self.keywords = keywords
        self.tokenizer = tokenizer
        self.start_len = input_ids.shape[1]
        self.keyword_ids = []
        for kw in keywords:
            ids = tokenizer(kw).input_ids
            if len(ids) > 1 and ids[0] == tokenizer.bos_token_id:
                ids = ids[1:]
            self.keyword_ids.append(torch.tensor(ids))
        self.max_keyword_len = max((len(k) for k in self.keyword_ids), default=0)

    def call_for_batch(self, output_ids, scores, **kwargs):
        offset = min(output_ids.shape[1] - self.start_len, self.max_keyword_len)
        for kid in self.keyword_ids:
            kid = kid.to(output_ids.device)
            if output_ids.shape[1] >= kid.shape[0] and (output_ids[0, -kid.shape[0]:] == kid).all():
                return True
        text = self.tokenizer.batch_decode(output_ids[:, -offset:], skip_special_tokens=True)[0]
        return any(kw in text for kw in self.keywords)

    def __call__(self, output_ids, scores, **kwargs):
        return all(self.call_for_batch(output_ids[i].unsqueeze(0), scores) for i in range(output_ids.shape[0]))
