cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== FULL"; python3 tools/conv_bench.py 2>&1 | grep -E "k3 s1"
for v in $VARS; do echo "== $v"; HAPPYPOSE_AMD_LIB=$PWD/happypose_amd/lib/abl/$v.so python3 tools/conv_bench.py 2>&1 | grep -E "k3 s1"; done
