for t in default 128x128 128x64; do
echo "== tile=$t"
if [ $t = default ]; then python tools/microbench.py bf16 s2a 2>&1 | grep -E "fwd|dX"; python tools/microbench.py bf16 s1a 2>&1 | grep -E "fwd|dX";
else FOCAL_GEMM_TILE=$t python tools/microbench.py bf16 s2a 2>&1 | grep -E "fwd|dX"; FOCAL_GEMM_TILE=$t python tools/microbench.py bf16 s1a 2>&1 | grep -E "fwd|dX"; fi
done
echo "== dW wide"
FOCAL_GEMM_DW_WIDE=1 python tools/microbench.py bf16 s2a 2>&1 | grep -E "dW"; FOCAL_GEMM_DW_WIDE=1 python tools/microbench.py bf16 s1a 2>&1 | grep -E "dW"
