OUT=gpurun_out/r05o; mkdir -p $OUT
for i in 1 2; do tools/class_latency; done | tee $OUT/class_latency.json
python -m pytest tests/test_hm_dropin.py tests/test_abi_cpu.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/tests.txt
python -c "
import sys; sys.path.insert(0,'hm-opencl_amd')
from hmme import api
e=api.Engine(0,64); print(e.device_info); e.close()"
