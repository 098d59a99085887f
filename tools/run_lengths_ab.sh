#!/bin/bash
# the lone call with the lean kernel's self-prepare / self-fold (GPU box, repo root): SYNTHHIP_SELF = 0 off (default), 1 both, 2 the fold, 3 the records
for m in 0 1 2 3; do echo "== SYNTHHIP_SELF=$m"; SYNTHHIP_SELF=$m python tools/probe.py run-lengths 2>&1 | head -7; done
