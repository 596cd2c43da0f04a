V="PAULE_HIP_FUSED=0/PAULE_HIP_FUSED=1,PAULE_HIP_FUSED_MIN_B=1/PAULE_HIP_FUSED=3,PAULE_HIP_FUSED_MIN_B=1"
for b in 33 64 96 128 160 192; do echo "## B=$b T=300"; AB_BATCH=$b timeout -k 10 200 python3 tools/ab_bench.py "$V" 3 10 2>&1 | grep -E 'median|Error|rror' ; done
for b in 32 64; do echo "## B=$b T=2000"; AB_BATCH=$b AB_FRAMES=2000 timeout -k 10 300 python3 tools/ab_bench.py "$V" 2 4 2>&1 | grep -E 'median|Error|rror' ; done
