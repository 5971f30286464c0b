#!/bin/bash
# Same-box A/B of build flags on the N > 1 step driven by one rank (tools/check_exchange.py --time): tools/ab_exchange.sh "<flags A>" "<flags B>" ...
for f in "$@"; do
  export IGS_EXTRA_FLAGS="$f"
  python -c "import igs_amd.build as b; b.build()" || exit 1
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/check_exchange.py --time 2>&1 | grep -E "^exchange|EXCHANGE_CHECK" | sed "s/^/[$f] /"
done
export IGS_EXTRA_FLAGS=""; python -c "import igs_amd.build as b; b.build()" > /dev/null 2>&1
