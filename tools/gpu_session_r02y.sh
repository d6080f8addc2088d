#!/bin/bash
# what the driver runs at round end, in its order: smoke(), then the default bench line
set +e
export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python bench.py 2>&1 | tail -1 | cut -c1-2500
