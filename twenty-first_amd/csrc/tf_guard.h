// tf_guard.h -- no C++ exception crosses the C ABI.
// The library is ordinary C++17 on the host side: std::vector / std::map / std::string for its tables, caches and messages, std::thread for
// the workers of tf_*_multi.  Any of them can throw (std::bad_alloc, std::length_error, std::system_error); an exception that leaves an
// extern "C" function is undefined behaviour for a C or Rust caller -- in practice std::terminate.  Every status-returning entry point of
// include/tf_hip.h is therefore a function-try-block
//     int tf_xxx(...) try { ... } TF_ABI_CATCH
// that turns what escapes into a status: TF_ERR_OUT_OF_MEMORY for a failed host allocation, TF_ERR_INTERNAL for anything else, the
// message kept for tf_last_error().  RAII owns everything the bodies hold (locks, DevBuf, scratch blocks), so unwinding releases it.
// This header depends on nothing but the status codes, so tests/test_abi_and_host.py can compile the mechanism with g++ alone.
#pragma once
#include <exception>
#include <new>
#include <stdexcept>

namespace tfi {
// records the message for tf_last_error() (never throws: a message that cannot be stored is dropped) and returns `status`
int abi_caught(const char* what, int status) noexcept;
}  // namespace tfi

#define TF_ABI_CATCH                                                                                                              \
    catch (const std::bad_alloc&) { return ::tfi::abi_caught("host allocation failed (std::bad_alloc)", TF_ERR_OUT_OF_MEMORY); }   \
    catch (const std::length_error& e_) { return ::tfi::abi_caught(e_.what(), TF_ERR_OUT_OF_MEMORY); }                              \
    catch (const std::exception& e_) { return ::tfi::abi_caught(e_.what(), TF_ERR_INTERNAL); }                                      \
    catch (...) { return ::tfi::abi_caught("unknown C++ exception", TF_ERR_INTERNAL); }
