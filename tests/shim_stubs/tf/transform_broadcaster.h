#pragma once
#include <ros/ros.h>
#include <geometry_msgs/Pose.h>
namespace tf {
struct Quaternion { double x, y, z, w; };
struct Matrix3x3 { explicit Matrix3x3(const Quaternion &) {} void getRPY(double &, double &, double &) const {} };
inline Quaternion createQuaternionFromRPY(double, double, double) { return Quaternion(); }
inline void quaternionMsgToTF(const geometry_msgs::Quaternion &, Quaternion &) {}
inline void quaternionTFToMsg(const Quaternion &, geometry_msgs::Quaternion &) {}
}
using tf::quaternionMsgToTF; using tf::quaternionTFToMsg;
