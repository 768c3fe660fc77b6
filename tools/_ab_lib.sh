cp on_device_image_captioning_amd/libodic_hip.so /tmp/lib_new.so
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "lowp or fp8 or f16" 2>&1 | tail -2
for i in 1 2; do
  for v in new old; do
    if [ $v = new ]; then cp /tmp/lib_new.so on_device_image_captioning_amd/libodic_hip.so; else cp tools/_build/libodic_lowp_old.so on_device_image_captioning_amd/libodic_hip.so; fi
    python bench.py --workload fp8b64 --no-roofline --no-parity --no-fp32 --no-exact --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('$v', j['value'], j['ms_per_step'])"
  done
done
cp /tmp/lib_new.so on_device_image_captioning_amd/libodic_hip.so
