for lib in "" orthosfm_amd/lib/exp/lib_noscore.so; do
  OSFM_HIP_LIBRARY=${lib:+$PWD/$lib} python bench.py --no-ba --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('${lib:-default}', round(d['ms_per_step'],2), round(d['with_geometric_verification']['ms_per_step'],2))"
done
