// hip_context.h -- which GPU the signature-preserving host classes (ORBextractor, ORBmatcher, Optimizer) run on, and the device
// context they share per calling thread.  The reference's classes have no notion of a device, so the choice is made out of band:
//   ORB_SLAM3::hip::SetDevice(n)   before the first call (e.g. from main(), one process per GPU: n = the rank's local GPU), else
//   the environment variable ORBHIP_DEVICE, else device 0.
// ORBmatcher objects are stack temporaries in Tracking, LocalMapping and LoopClosing, which run concurrently (SURVEY 8b "Threading"):
// every calling thread gets ONE orbhip_ctx of its own (own HIP stream, own scratch arena), created on first use and destroyed when the
// thread ends.  Teardown order: a thread's context is a thread_local object -- for worker threads it is released at thread exit; for the
// main thread C++ runs thread_local destructors BEFORE static destructors and atexit handlers (basic.start.term), i.e. while the HIP
// runtime (a shared library with static state) is still alive.  Threads still running at process exit never run their destructors: their
// contexts are left to the driver's process teardown, nothing calls into HIP from an exit handler.
#pragma once
#include "../../include/orbhip.h"

namespace ORB_SLAM3 {
namespace hip {
void SetDevice(int device);
int GetDevice();
// nullptr (after one message on stderr per thread) when no context can be created: there is no CPU fallback, callers return their
// "nothing found / nothing changed" result
orbhip_ctx *ThreadContext();
}  // namespace hip
}  // namespace ORB_SLAM3
