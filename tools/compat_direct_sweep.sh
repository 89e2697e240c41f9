# how many calls under way a call may find and still launch on its own (compat_bench's 11th argument)
for d in 4 8 12 16 24 32; do timeout -k 10 200 build/compat_bench 2 8,16,32,64,255 2 0 1 -1 -1 -1 1 $d | grep -v "host CPU\|per batch\|wake tree\|host threads\|small batches" || exit 1; done
