// rrt_internal.h -- helpers shared by the translation units of librrt_hip.so (not part of the ABI).
#ifndef RRT_INTERNAL_H
#define RRT_INTERNAL_H

#include <string>

namespace rrt {
// Records the message returned by rrt_last_error(NULL) on the calling thread.
void set_thread_error(const std::string& msg);
}  // namespace rrt
#endif
