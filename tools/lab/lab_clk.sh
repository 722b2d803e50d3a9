cd /root/repo
(timeout 200 python bench.py --steps 1200 --warmup 3 --no-cpu-baseline --no-kernel-timer > gpurun_out/r02n_bench.json 2>/dev/null &) 
sleep 30
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction|memory)" | head -8; echo ---; sleep 2; done
wait
tail -c 300 gpurun_out/r02n_bench.json
