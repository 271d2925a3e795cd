for rnd in 1 2; do
for lib in libtf_hip.so libtf_hip_x1.so libtf_hip_x2.so libtf_hip_x3.so libtf_hip_x4.so; do
  r=$(TF_HIP_LIBRARY=$PWD/twenty-first_amd/$lib python bench.py --no-extra --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'])")
  echo "round $rnd $lib: $r"
done; done
