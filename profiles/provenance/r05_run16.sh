timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_shim.py -m gpu -x -q -k "1024 or golden or random or phase" 2>&1 | tail -2
echo "== 1024^2 x 4"; N=1024 C=4 STEPS=1000 REPS=3 EXTRA="" tools/ab_4096.sh 2>&1 | grep -v amdgpu
echo "== 1024^2 x 4 fp16"; N=1024 C=4 STEPS=1000 REPS=2 EXTRA="--spectrum fp16" tools/ab_4096.sh 2>&1 | grep -v amdgpu
echo "== 1024^2 x 1"; N=1024 C=1 STEPS=2000 REPS=2 EXTRA="" tools/ab_4096.sh 2>&1 | grep -v amdgpu
echo "== 512^2 x 1"; N=512 C=1 STEPS=2000 REPS=2 EXTRA="" tools/ab_4096.sh 2>&1 | grep -v amdgpu
echo "== 512^2 x 4"; N=512 C=4 STEPS=2000 REPS=2 EXTRA="" tools/ab_4096.sh 2>&1 | grep -v amdgpu
echo "== 2048^2 x 1"; N=2048 C=1 STEPS=500 REPS=2 EXTRA="" tools/ab_4096.sh 2>&1 | grep -v amdgpu
