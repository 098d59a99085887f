cd "${GRAFT_REPO_ROOT:-/root/repo}"
LABEL=r02 SYNTHHIP_LIB=tools/ab/libsynthhip_r02.so python tools/dbg1.py 2>&1 | grep -v rerender
LABEL=guard SYNTHHIP_LIB=tools/ab/libsynthhip_guard.so python tools/dbg1.py 2>&1 | grep -v rerender
