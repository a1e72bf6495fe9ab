#!/usr/bin/env python3
"""``python run_nerf.py --config <experiment.yml>`` -- see nerf_kinematics_amd/run_nerf.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from nerf_kinematics_amd.run_nerf import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
