export TMPDIR=/tmp
python tools/host_gaps.py 12500 30 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"
python tools/host_gaps.py 100000 20 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"
