# (round 5) k_match_hc_sparse's ring: the product's size against other sizes (libgzpx_l<entries>.so, -DGZPX_HS_LIST=...)
# and, if present, the build before (libgzpx_prev.so): the texts of tools/exp_text_seeds.py at levels 3 and 4 and the bench
# slab through bench.py.     tools/gpu_r5_ring.sh <outdir> <lib> [<lib> ...]
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift; mkdir -p $O
LIBS="$@"
for LV in 3 4; do
  timeout 200 python tools/exp_text_seeds.py $LV $LIBS 2>&1 | grep level | tee $O/seeds_l$LV.txt
done
for L in gzp_amd/lib/libgzpx.so $LIBS; do
  for LV in 3 4; do
  timeout 100 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --workload bgzf3 --level $LV --lib $L > $O/b.json 2> $O/b.err
  python - <<PY
import json
d = json.loads(open("$O/b.json").read().strip().splitlines()[-1])
print("$L level $LV:", d["ms_per_step"], "ms; match+parse", d["roofline"]["stage_ms"]["k_match_hc+k_parse_hc"], d["config"]["stream_sha256"][:10])
PY
  done
done
