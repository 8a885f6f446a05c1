#!/bin/bash
# round 5, call H: this library's GEMM kernels beside the vendor's (torch F.linear = hipBLASLt / rocBLAS) on every GEMM shape of the step, plain C = A W^T;
# a rocprofv3 kernel trace of the same script names the vendor kernels (their macro-tile shapes).
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05h; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 600 python scripts/vendor_gemm_ref.py 3 2>&1 | grep -v amdgpu.ids | tee $O/vendor_gemm_ref.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o v -- python $R/scripts/vendor_gemm_ref.py 1 > $R/$O/trace.log 2>&1 )
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/vendor_kernel_stats.csv \;
rm -rf $O/trace
cut -c1-200 $O/vendor_kernel_stats.csv | head -30
