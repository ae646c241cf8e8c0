# shader clock / power the chip sustains under the step: rocm-smi sampled while bench.py runs (one sample per ~0.3 s)
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/clock_samples.txt &
SAMPLER=$!
python bench.py --no-cpu-baseline --no-optimizer --no-clock-probe --profile-steps 0 --steps 60 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
wait $SAMPLER
grep -c . gpurun_out/clock_samples.txt
sort gpurun_out/clock_samples.txt | uniq -c | sort -rn | head -20
