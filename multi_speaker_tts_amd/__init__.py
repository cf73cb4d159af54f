"""MI355X-native hot path of CODEJIN/multi_speaker_tts."""
import os as _os

# The decoder loops are chains of ~6 400 dependent launches per train step: every kernel's first instruction is the fetch of its
# argument block.  With the blocks in device memory (the HIP runtime's default on this GPU) that fetch costs what any cold read
# costs; with HIP_FORCE_DEV_KERNARG=0 they sit in host memory and the step goes from 87.8 to 100.5 ms (tools/env_ab.sh,
# tools/kernarg_probe.hip).  Pin the default before the HIP runtime reads it; an explicit setting of the caller wins.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
# Multi-process GPU work (RCCL, tensors shared between ranks) needs dmabuf IPC on this driver stack; the variable is read when the
# HSA runtime comes up, i.e. at the first torch.cuda call of the process - setting it inside init_process_group is too late for a
# caller that has already picked its device.  An explicit setting of the caller wins.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
