# Per-pair time of the aggregation's forms over batch sizes (Teddy size, D = 128).
cd $GRAFT_REPO_ROOT
for B in 1 2 4 8 16 32 64 128; do
  echo "== B=$B"; FORMS=${FORMS:-walks,tiled,skew} timeout -k 10 200 python tools/sgm_forms.py ${SHAPE:-375 450 128} $B 5 2>&1 | grep -E "fused " | grep -v unfused
done
