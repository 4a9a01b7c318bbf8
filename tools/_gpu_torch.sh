mkdir -p gpurun_out
(python -c "import torch; print('plain', torch.cuda.is_available(), torch.cuda.device_count())"
 python -c "
import sys
sys.path.insert(0, 'rust-brotli_amd')
import brotli_mi355x as bm
lib = bm.default_library()
print('lib', lib.device_name(), len(lib.compress(b'hello hello hello hello', 5, 22)))
import torch
print('after lib', torch.cuda.is_available(), torch.cuda.device_count())
"
 env | grep -i "hip\|rocr\|hsa\|cuda\|gpu" 
 timeout 100 python -m pytest tests/test_multi_gpu_plumbing.py -x -q -m gpu 2>&1 | tail -5) > gpurun_out/r04_torch_check.log 2>&1
cat gpurun_out/r04_torch_check.log | tail -30
