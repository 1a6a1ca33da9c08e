"""MI355X-native rover env.step() hot path behind the reference's RLTask / RoverTask API.

Reference: abmoRobotics/isaac_rover_2.0, ``omniisaacgymenvs/tasks/rover.py`` (task),
``tasks/base/rl_task.py`` (buffers + post_physics_step).  The arithmetic lives in
``csrc/`` (hand-written HIP for gfx950 behind the C ABI of ``include/rover_step.h``).
"""
__all__ = ["synth"]
