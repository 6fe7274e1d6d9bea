for v in ${VARS:-0 8 12}; do echo VAR=$v; FW_GEMM_TILE=256 FW_GEMM_VAR=$v timeout 300 python tools/microbench.py --iters 3 --only gemm 2>&1 | grep -E "dit o/q|ffn2|vggt fc1"; done
