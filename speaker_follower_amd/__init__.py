"""MI355X-native hot path of ronghanghu/speaker_follower (see README.md / DESIGN.md)."""
import os as _os

# Kernel arguments in device memory (a HIP runtime switch read when the runtime initialises, i.e. at the first HIP call of
# the process -- importing this package before torch touches the GPU is enough): measured on MI355X in round 6, the
# launch-by-launch issue of a rollout's ~130 kernels takes 1.69 ms of kernel time with it and 1.98 ms without (the command
# processor fetches the arguments from HBM instead of host memory); graph replays gain 0.5 %.  `setdefault`: an explicit
# setting of the user wins.
_os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
