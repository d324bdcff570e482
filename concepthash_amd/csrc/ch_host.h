// Host-side error plumbing of libconcepthash_hip, usable from plain C++ translation units (no HIP headers): the last-error string behind
// ch_last_error() and the argument check every C-ABI entry point opens with.  ch_common.h (device helpers + the HIP-status macros) includes it.
#pragma once
#include <stdint.h>

#include <string>

void ch_set_error(const std::string &msg);
#define CH_REQUIRE(cond, msg)                                                                                \
    do {                                                                                                     \
        if (!(cond)) {                                                                                       \
            ch_set_error(std::string("invalid argument: ") + (msg));                                         \
            return 2;                                                                                        \
        }                                                                                                    \
    } while (0)
