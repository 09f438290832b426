"""navbot_ppo_amd -- MI355X-native batched LiDAR-navigation simulator + PPO hot path.

Drop-in for the reference's ``project_ppo`` ``Env.reset()/Env.step()`` surface
(project_ppo/src/environment_new.py:26-382) and the rollout / return-scan / PPO-update loop
that consumes it (project_ppo/src/ppo.py:218-737).  The env step and the return scan are HIP
kernels for gfx950 behind the C ABI in include/navsim.h (libnavsim.so); there is no CPU path.
"""
__version__ = "0.1.0"
