export NHIP_TUNABLES=1  # (the library reads its environment switches only then)
for q in 1 2 4 8; do for o in 0 1; do echo "queues=$q lpt=$o: $(NHIP_BNB_QUEUES=$q NHIP_QUICK_ORDER=$o timeout -k 5 100 python tools/bnb_quick.py 2>/dev/null | tr '\n' ' ')"; done; done
