#pragma once
#include <string>
namespace std_msgs { struct Header { unsigned seq; double stamp; std::string frame_id; }; }
