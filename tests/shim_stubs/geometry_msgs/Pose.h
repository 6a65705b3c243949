#pragma once
namespace geometry_msgs { struct Quaternion { double x, y, z, w; }; struct Point { double x, y, z; };
struct Pose { Point position; Quaternion orientation; }; }
