cd /root/repo; mkdir -p gpurun_out/r6e
timeout 2400 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -30 > gpurun_out/r6e/suite.txt; tail -6 gpurun_out/r6e/suite.txt
for i in 1 2 3; do timeout 600 python3 bench.py --no-cpu-baseline --steps 400 > gpurun_out/r6e/b1m_$i.json 2>/dev/null; python3 tools/pj.py < gpurun_out/r6e/b1m_$i.json; done
timeout 600 python3 bench.py --no-cpu-baseline --workload sparse-200k --steps 400 > gpurun_out/r6e/b200k.json 2>/dev/null; python3 tools/pj.py < gpurun_out/r6e/b200k.json
timeout 600 python3 bench.py --no-cpu-baseline --workload dense-50k --steps 30 > gpurun_out/r6e/bdense.json 2>/dev/null; python3 tools/pj.py < gpurun_out/r6e/bdense.json
timeout 600 python3 bench.py --no-cpu-baseline --workload sparse-5m --steps 10 --warmup 3 > gpurun_out/r6e/b5m.json 2>/dev/null; python3 tools/pj.py < gpurun_out/r6e/b5m.json
