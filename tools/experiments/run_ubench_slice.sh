#!/bin/bash
# on the GPU box: build and run the bit-slice microbenchmark
cd "$(dirname "$0")"
python3 gen_csa.py 60 CSA60 > csa60.inc 2>/dev/null && python3 gen_csa.py 30 CSA30 > csa30.inc 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 ubench_slice.hip -o /tmp/ubench_slice || exit 1
/tmp/ubench_slice "$@"
