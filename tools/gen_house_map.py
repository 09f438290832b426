#!/usr/bin/env python3
"""Build-container tool: converts the reference's turtlebot3_house SDF (read from /root/reference at run time) into the
segment list navbot_ppo_amd/assets/turtlebot3_house_segments.npy with navbot_ppo_amd.sdf_ingest.  The asset is derived
geometry (numbers), the SDF itself is not copied."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from navbot_ppo_amd import sdf_ingest
src = "/root/reference/turtlebot3_simulations/turtlebot3_gazebo/models/turtlebot3_house/model.sdf"
seg, st = sdf_ingest.sdf_to_segments(src, scan_z=0.182, cylinder_sides=12)
out = os.path.join(R, "navbot_ppo_amd", "assets", "turtlebot3_house_segments.npy")
np.save(out, seg)
print(out, st)
