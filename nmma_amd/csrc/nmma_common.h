// nmma_common.h -- what the translation units of libnmma_hip.so share on the host side.
#pragma once
#include <string>

namespace nmma {
// records the message nmma_last_error() returns (thread-local, defined in em_api.inc) and returns 1
int fail(const std::string& msg);
}  // namespace nmma
