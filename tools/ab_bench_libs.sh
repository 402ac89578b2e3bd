for rep in 1 2 3; do
  for name in "" plainoff; do
    lib=${name:+$PWD/tools/_build/libmanytor_hip_$name.so}
    echo -n "${name:-default} "
    MT_LIB_OVERRIDE=$lib python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"
  done
done
