for i in 1 2 3; do
echo -n "old "; MPSR_LIB_PATH=abl/libold.so MPSR_PARTIAL_LIB=1 python tools/train_bench.py --steps 12 --warmup 3 --dgrad-bank 0 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
echo -n "new "; python tools/train_bench.py --steps 12 --warmup 3 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
done
