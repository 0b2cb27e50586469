export TMPDIR=/tmp
echo "== fuzz_parity 160 cases seed 606"; timeout 1500 python tools/fuzz_parity.py 160 606 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -6
echo "== fuzz_parity warm 100 seed 607"; timeout 1500 python tools/fuzz_parity.py 100 607 warm 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -6
echo "== fuzz_parity long 30 seed 608"; timeout 1500 python tools/fuzz_parity.py 30 608 long 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -6
echo "== fuzz_em 80 seed 61"; timeout 1500 python tools/fuzz_em.py 80 61 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -6
for c in tools/cases/*.npz; do echo "== $c"; timeout 300 python tools/fuzz_case.py $c 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | grep "STM_DEBUG_FLAGS" ; done
