echo "== 1024^2 x 4"; N=1024 C=4 STEPS=1000 REPS=3 EXTRA="" tools/ab_4096.sh 2>&1 | grep -v amdgpu
echo "== 512^2 x 1"; N=512 C=1 STEPS=2000 REPS=3 EXTRA="" tools/ab_4096.sh 2>&1 | grep -v amdgpu
echo "== 512^2 x 4"; N=512 C=4 STEPS=1000 REPS=2 EXTRA="" tools/ab_4096.sh 2>&1 | grep -v amdgpu
echo "== 1024^2 x 16"; N=1024 C=16 STEPS=200 REPS=2 EXTRA="" tools/ab_4096.sh 2>&1 | grep -v amdgpu
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "512 or 1024 or golden or random or phase" 2>&1 | tail -1
DATUM_OCEAN_HIP_LIB=$(realpath datum_amd/lib/variants/lib_e16wave.so) timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "512 or 1024 or golden or random or phase" 2>&1 | tail -1
