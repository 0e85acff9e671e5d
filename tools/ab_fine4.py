"""A/B of the four-channel GroupNorm partial sums from the limb epilogues (128-channel tensors): `off` restores round 4's
rule (8-channel sums only: 128-channel tensors take a statistics pass).  python tools/ab_fine4.py on|off <bench args...>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv.pop(1)
if mode == "off":
    from psld_amd import ops
    orig = ops.gn_part_supported.__wrapped__

    def only8(b, hw, c):
        return orig(b, hw, c) and (c // ops.gn_groups(c)) % 8 == 0
    ops.gn_part_supported = only8
import bench  # noqa: E402

sys.exit(bench.main())
