echo "== 4096^2 fp16"; N=4096 C=1 STEPS=200 REPS=3 EXTRA="--spectrum fp16" tools/ab_4096.sh 2>&1 | grep -v amdgpu
echo "== 4096^2 fp32"; N=4096 C=1 STEPS=200 REPS=3 EXTRA="" tools/ab_4096.sh 2>&1 | grep -v amdgpu
echo "== 2048^2 x 1"; N=2048 C=1 STEPS=500 REPS=3 EXTRA="" tools/ab_4096.sh 2>&1 | grep -v amdgpu
echo "== 2048^2 x 4"; N=2048 C=4 STEPS=200 REPS=2 EXTRA="" tools/ab_4096.sh 2>&1 | grep -v amdgpu
DATUM_OCEAN_HIP_LIB=$(realpath datum_amd/lib/variants/lib_packedall.so) timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "2048 or 4096" 2>&1 | tail -1
