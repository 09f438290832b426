"""Stub modules that let the reference's pure-arithmetic code import without ROS/Gazebo.

Used ONLY by tests/golden/generate_golden.py, in the build container, to run the
reference (read from /root/reference at run time, never copied) and record golden
input/output vectors.  Nothing here is reference code: these are empty stand-ins for
`rospy`, `roslaunch`, the ROS message packages, `gym`, `tensorboardX` and
`torchvision`, none of which is installed in the image.
"""
import sys
import types


class _Anything:
    """Accepts any construction / call / attribute access and does nothing."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        return _Anything()


class _XYZ:
    def __init__(self):
        self.x = 0.0
        self.y = 0.0
        self.z = 0.0
        self.w = 1.0


class Pose:
    def __init__(self):
        self.position = _XYZ()
        self.orientation = _XYZ()
        # the reference reads `self.position.x` right after `self.position = Pose()`
        self.x = 0.0
        self.y = 0.0


class Twist:
    def __init__(self):
        self.linear = _XYZ()
        self.angular = _XYZ()


class _Time:
    @staticmethod
    def now():
        return 0.0


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install():
    class ServiceException(Exception):
        pass

    rospy = _mod(
        "rospy",
        Publisher=_Anything,
        Subscriber=_Anything,
        ServiceProxy=_Anything,
        ServiceException=ServiceException,
        wait_for_service=lambda *a, **k: None,
        wait_for_message=lambda *a, **k: None,  # patched per call by the generator
        init_node=lambda *a, **k: None,
        is_shutdown=lambda: False,
        sleep=lambda *a, **k: None,
        Time=_Time,
        Duration=lambda *a, **k: 0.0,
    )
    _mod("roslaunch")
    _mod("geometry_msgs")
    _mod("geometry_msgs.msg", Twist=Twist, Point=_XYZ, Pose=Pose)
    _mod("sensor_msgs")
    _mod("sensor_msgs.msg", LaserScan=_Anything, Image=_Anything)
    _mod("nav_msgs")
    _mod("nav_msgs.msg", Odometry=_Anything)
    _mod("std_srvs")
    _mod("std_srvs.srv", Empty=_Anything)
    _mod("gazebo_msgs")
    _mod("gazebo_msgs.srv", SpawnModel=_Anything, DeleteModel=_Anything, GetModelState=_Anything)
    # ppo.py / nets
    _mod("gym")
    _mod("tensorboardX", SummaryWriter=_Anything)
    tv = _mod("torchvision")
    tv.models = _mod("torchvision.models")
    tv.transforms = _mod("torchvision.transforms")
    return rospy
