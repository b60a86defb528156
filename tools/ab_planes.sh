# planes pipeline on / off (and fp16x2) at several batch sizes and at cfg4 (run on the GPU box from the repo root)
cd $GRAFT_REPO_ROOT
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for b in 8192 16384 32768 65536; do
  USFLOWS_AMD_PLANES=0 python bench.py --no-cpu-baseline --no-fast-mode --no-kernel-timing --batch $b --steps 20 2>/dev/null | p "B=$b bf16x3 fp32-activations"
  USFLOWS_AMD_PLANES=1 python bench.py --no-cpu-baseline --no-fast-mode --no-kernel-timing --batch $b --steps 20 2>/dev/null | p "B=$b bf16x3 planes"
  USFLOWS_AMD_PLANES=1 python bench.py --no-cpu-baseline --no-fast-mode --no-kernel-timing --batch $b --steps 20 --gemm f16x2 2>/dev/null | p "B=$b f16x2 planes"
done
USFLOWS_AMD_PLANES=0 python bench.py --config cfg4 --no-cpu-baseline --no-kernel-timing --steps 3 --warmup 1 2>/dev/null | p "cfg4 bf16x3 fp32-activations"
USFLOWS_AMD_PLANES=1 python bench.py --config cfg4 --no-cpu-baseline --no-kernel-timing --steps 3 --warmup 1 2>/dev/null | p "cfg4 bf16x3 planes"
python bench.py --config cfg4 --no-cpu-baseline --no-kernel-timing --steps 3 --warmup 1 --gemm f16x2 2>/dev/null | p "cfg4 f16x2 planes"
USFLOWS_AMD_PLANES=0 python bench.py --no-cpu-baseline --no-fast-mode --no-kernel-timing --conj --householder 1 --steps 10 2>/dev/null | p "conj+hh1 bf16x3 fp32-activations"
USFLOWS_AMD_PLANES=1 python bench.py --no-cpu-baseline --no-fast-mode --no-kernel-timing --conj --householder 1 --steps 10 2>/dev/null | p "conj+hh1 bf16x3 planes"
python bench.py --no-cpu-baseline --no-fast-mode --no-kernel-timing --conj --householder 1 --steps 10 --gemm f16x2 2>/dev/null | p "conj+hh1 f16x2 planes"
