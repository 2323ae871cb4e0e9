for S in l3c2 big qkv ffn2 l2c2; do
for W in 0 1; do
  echo -n "shape $S WIDE=$W: "
  AVCER_GEMM_WIDE=$W python tools/gemm_one.py $S x3s 10 2>&1 | grep "TF/s" | grep -v "===" | sed "s/  */ /g" | cut -d" " -f1-14
done; done
