#!/bin/bash
# Run ON THE GPU BOX: smoother batch-of-256 timings with the streamed-weights message kernel off / on (tools/latency.py lines)
R=${GRAFT_REPO_ROOT:-$PWD}
for s in 0 1; do echo "GNNMP_SM_STREAM=$s"; GNNMP_SM_STREAM=$s python $R/tools/latency.py --smoother-only 2>&1 | grep "^smooth"; done
