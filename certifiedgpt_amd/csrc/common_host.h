// common_host.h -- error plumbing shared by the C-ABI translation units.
#pragma once
#include <string>
#include "../../include/cgpt.h"

// Records `msg` as the thread's last error and returns `code` (never throws across the ABI).
cgpt_status cgpt_fail(cgpt_status code, const std::string& msg);
