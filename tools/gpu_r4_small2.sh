cd "$GRAFT_REPO_ROOT"; for spec in "10000 10 0.0" "1294 50 0.85"; do timeout 300 python tools/latency_breakdown.py $spec 2>&1 | grep -v "^$" | head -12; done
