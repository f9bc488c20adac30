set -u
R=$GRAFT_REPO_ROOT
cd $R
export HOME=/tmp/freshhome; mkdir -p $HOME       # an empty user cache: only the in-tree seeds can help
ls spartan_amd/csrc/jit_seed | wc -l
SP_JIT_VERBOSE=1 python bench.py --steps 2 --warmup 1 --only hbm 2> gpurun_out/seed_err.txt | python -c "
import sys,json; d=json.loads(sys.stdin.read())['hbm']; print({k:v for k,v in d.items() if 'chain' in k or 'sq_dev' in k or 'f64' in k or 'copy' in k})"
grep -c "loaded" gpurun_out/seed_err.txt; grep -c " compiled " gpurun_out/seed_err.txt; grep "jit\]" gpurun_out/seed_err.txt | head -12
timeout 900 python -m pytest tests -m gpu -q -n 4 2>&1 | tail -3
