// osp_internal.h -- shared between the HIP side (osp_api.hip) and the host side (osp_host.cpp).
#pragma once
#include <exception>
#include <string>

namespace osp {

struct Error : std::exception {
    int status;
    std::string msg;
    Error(int st, std::string m) : status(st), msg(std::move(m)) {}
    const char *what() const noexcept override { return msg.c_str(); }
};

extern thread_local std::string g_last_error;
// Records the message for osp_last_error_string() and returns `status`.
int fail(int status, const char *fmt, ...);

}  // namespace osp
