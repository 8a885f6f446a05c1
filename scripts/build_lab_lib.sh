#!/bin/bash
# builds videollama2_amd/libvl2hip_lab.so = the product sources with -DVL2_LAB (the experiments that were measured and lost: gemm8, the
# issue-order / stamped forms of gemm9, the woven LDS-DMA issue, the two-accumulator persistent GEMM, stream-K, the decode tail engine,
# attention + elected combine in one launch).  Loaded by scripts and tests through videollama2_amd._lib.set_lab(True); never by the product.
cd "$(dirname "$0")/.." && exec python -m videollama2_amd.csrc.build --lab "$@"
