// Internal glue between translation units (not part of the C ABI).
#pragma once
namespace tg {
// Records `msg` as this thread's tg_last_error() and returns `code` (defined in tg_api.hip).
int report_error(int code, const char* msg);
}  // namespace tg
