import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import whisprrec_amd.rotating as R
R.RotatingBprmf._await_part = lambda self, part: self._ready.pop(part, None)      # MUTATION: never wait for an incoming part
import test_hip_rotating_loopback as T
bad = 0
for args in ((2, 2, 2), (3, 2, 64), (4, 3, 1)):
    try:
        T.test_virtual_ranks_epoch_equals_single_process(*args); print(args, "passed (mutation not detected)")
    except AssertionError as e:
        bad += 1; print(args, "FAILED as it should:", str(e)[:80])
print("detected", bad, "of 3")
