for d in 0 1 2 4 8 6 7 15; do echo -n "dbg=$d "; GNNMP_MP_DBG=$d tools/cfg3.sh "$@"; done
