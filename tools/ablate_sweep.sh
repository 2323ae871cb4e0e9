for S in l3c2 big qkv; do
for AB in 0 8; do
  echo -n "shape $S ablate=$AB: "
  AVCER_GEMM_ABLATE=$AB python tools/gemm_one.py $S x3s 10 2>&1 | grep "TF/s" | grep -v "===" | sed "s/  */ /g" | cut -d" " -f1-14
done; done
