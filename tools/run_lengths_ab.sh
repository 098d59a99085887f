for m in 0 1 2 3; do echo "== SYNTHHIP_NO_SELF=$m"; SYNTHHIP_NO_SELF=$m python tools/probe.py run-lengths 2>&1 | head -7; done
