for w in "modeG" "prime_pin modeG" "prime_copy modeG" "prime_stage modeG"; do echo "== $w"; python3 scripts/pcie_bisect.py $w 2>&1 | tail -1; done
