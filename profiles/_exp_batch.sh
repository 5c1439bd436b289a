# SA2 scale batch A/B: the switch is "set = off": compare unset vs a build-time toggle through PRCNN_NO_SA2_BATCH
VALS="unset 1" bash profiles/_exp_ab.sh PRCNN_NO_SA2_BATCH
