cd ${GRAFT_REPO_ROOT:-/root/repo}
(AMD_LOG_LEVEL=3 timeout 1500 python tools/reach_sweep.py 2>&1 | grep -o "ShaderName : .*\|Error.*\|Traceback.*" | grep "sk::\|Error\|Traceback" | sort | uniq -c | sort -rn > gpurun_out/r05_reach_names.txt)
wc -l gpurun_out/r05_reach_names.txt; grep -c "Error\|Traceback" gpurun_out/r05_reach_names.txt
timeout 3000 python -m pytest tests/ -q -m gpu 2>&1 | tail -2
