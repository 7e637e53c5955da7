cd $GRAFT_REPO_ROOT
echo "mega auto (B=1 -> persistent kernel)"; python tools/online_probe.py 1000 2>&1 | grep "graph="
echo "mega off"; RIP_ENCODER_MEGA=0 python tools/online_probe.py 1000 2>&1 | grep "graph="
