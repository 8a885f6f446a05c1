"""Oracle shim: `imageio` is imported at the top of /root/reference/videollama2/mm_utils.py:8-13 but is not in
this image.  Only the file-path branches of process_video need it; the ndarray / PIL branches do not."""


def _missing(*a, **k):
    raise ImportError("imageio is not available in this image (oracle shim)")


VideoReader = cpu = get_reader = cvtColor = _missing
COLOR_RGBA2RGB = 0
