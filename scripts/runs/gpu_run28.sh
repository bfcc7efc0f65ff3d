python3 scripts/pcie_bisect.py first after_headline modeG 2>&1 | tail -6
python3 scripts/pcie_bisect.py modeG 2>&1 | tail -3
