cd $GRAFT_REPO_ROOT
ANNCHOR_HIP_LIB=$PWD/tools/_variants/libemd_prof.so timeout 300 python tools/c4_time.py 2>&1 | tail -19
