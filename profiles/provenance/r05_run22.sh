./tools/dbg/bin/stamps_4096h 2>&1 | head -60 > gpurun_out/stamps_4096h.txt
echo "== 4096^2 fp16"; N=4096 C=1 STEPS=200 REPS=3 EXTRA="--spectrum fp16" tools/ab_4096.sh 2>&1 | grep -v amdgpu
echo "== 4096^2 fp32"; N=4096 C=1 STEPS=200 REPS=2 EXTRA="" tools/ab_4096.sh 2>&1 | grep -v amdgpu
echo "== 2048^2 x 1"; N=2048 C=1 STEPS=500 REPS=2 EXTRA="" tools/ab_4096.sh 2>&1 | grep -v amdgpu
echo "== 1024^2 x 4"; N=1024 C=4 STEPS=1000 REPS=3 EXTRA="" tools/ab_4096.sh 2>&1 | grep -v amdgpu
echo "== 512^2 x 1"; N=512 C=1 STEPS=2000 REPS=2 EXTRA="" tools/ab_4096.sh 2>&1 | grep -v amdgpu
