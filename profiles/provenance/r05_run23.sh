# extended seeded differential runs on the final build (beyond the freeze's): more seeds, more cases
{
for seed in 31 32 33 34; do timeout 900 python tools/dbg/fuzz.py 250 $seed 2>&1 | tail -1; done
for seed in 35 36; do FUZZ_SIZES=2048,4096 timeout 1200 python tools/dbg/fuzz.py 20 $seed 2>&1 | tail -1; done
echo "-- api_fuzz"; for seed in 41 42 43 44; do timeout 1200 python tools/dbg/api_fuzz.py 120 $seed 60 2>&1 | tail -1; done
FUZZ_SIZES=1024,2048 timeout 1500 python tools/dbg/api_fuzz.py 12 45 30 2>&1 | tail -1
echo "-- host_fuzz"; for seed in 51 52 53 54; do timeout 1200 python tools/dbg/host_fuzz.py 100 $seed 150 2>&1 | tail -1; done
} > gpurun_out/fuzz_extended.txt 2>&1
tail -20 gpurun_out/fuzz_extended.txt | cut -c1-200
