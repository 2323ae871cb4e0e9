for S in l3c2 big qkv l1c3; do
for PR in 0 1; do
  echo -n "shape $S PRIO=$PR: "
  AVCER_GEMM_PRIO=$PR python tools/gemm_one.py $S x3s 10 2>&1 | grep "TF/s" | grep -v "===" | sed "s/  */ /g" | cut -d" " -f1-14
done; done
