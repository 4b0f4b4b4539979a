timeout 1500 python tools/fuzz_campaign.py 9000 100 2000 2>&1 | tail -2
for s in 21 22 23 24 25 26; do timeout 300 python tools/fuzz_big.py $s 2>&1 | tail -1; done
