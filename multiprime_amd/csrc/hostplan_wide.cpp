// hostplan_wide.cpp — part of libmprime_hip.so: the host stage of hostplan.cpp compiled a second time for primers of 33..63 bases
// (keys of four 64-bit words, 64-bit window words); every exported name carries the suffix _w64 and is reached through the entry
// points of hostplan.cpp only.
#define MP_PLAN_WIDE 1
#if defined(__GNUC__) && !defined(__clang__)
#pragma GCC diagnostic ignored "-Wsubobject-linkage"      // the plan type of an included file: one translation unit all the same
#endif
#include "hostplan.cpp"
