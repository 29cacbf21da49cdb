cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
READ_SMALL_PINNED=1 python tools/ubench/read_small.py 2560044 1024 2>&1 | grep -v amdgpu.ids
READ_SMALL_PINNED=1 python tools/ubench/read_small.py 643200 2048 2>&1 | grep -v amdgpu.ids
