#!/bin/bash
export R=r04
mkdir -p gpurun_out/r04i
python -m pytest tests -q -m gpu -v > gpurun_out/${R}_gpu_tests.txt 2>&1; grep -E "passed|failed" gpurun_out/${R}_gpu_tests.txt | tail -1
cp gpurun_out/parity_table.txt gpurun_out/${R}_parity_table.txt
{
for rep in 1 2 3; do
echo "-- shipped (nt, the compiler's builtin)"; python tools/gen_bench.py 64 1024
echo "-- plain stores"; DATUM_OCEAN_HIP_LIB=$(realpath datum_amd/lib/genvariants/lib_genst0.so) python tools/gen_bench.py 64 1024
echo "-- sc0 sc1 (asm + wait states)"; DATUM_OCEAN_HIP_LIB=$(realpath datum_amd/lib/genvariants/lib_genst17.so) python tools/gen_bench.py 64 1024
done
} > gpurun_out/r04i/gen_store_policy.txt 2>&1
grep -v libdrm gpurun_out/r04i/gen_store_policy.txt
bash tools/profile_gen.sh ${R}_prof_gen_64 64 > /dev/null 2>&1
bash tools/profile_gen.sh ${R}_prof_gen_1024 1024 > /dev/null 2>&1
for d in gen_64 gen_1024; do cp gpurun_out/${R}_prof_$d/summary.txt gpurun_out/${R}_summary_$d.txt; done
rm -rf gpurun_out/${R}_prof_*
python bench.py --steps 20 --warmup 5 > gpurun_out/r04i/bench20.json 2>/dev/null
