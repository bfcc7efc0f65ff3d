set -o pipefail
echo "== early alloc"; python3 scripts/trace_host.py after_resident close | tail -3 || exit 1
echo "== late alloc"; python3 scripts/trace_host.py after_resident close late_alloc | tail -3 || exit 1
echo "== late alloc, no resident phase"; python3 scripts/trace_host.py late_alloc | tail -3 || exit 1
lscpu | grep -i "numa\|socket\|model name" | head; nproc
timeout -k 10 300 python3 scripts/c3_two_pass.py 2>/dev/null || exit 1
