#pragma once
#include <cstdio>
#include <string>
namespace ros { namespace param { template <typename T> bool get(const std::string &, T &) { return false; } } }
#define ROS_INFO(...) ((void)0)
#define ROS_INFO_STREAM(x) ((void)0)
#define ROS_ERROR(...) ((void)0)
#define ROS_FATAL(...) ((void)0)
