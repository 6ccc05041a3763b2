// gbn_guard.hpp -- the exception firewall of the C ABI ("no exceptions may cross the C boundary").
// Every extern "C" entry point that can allocate or calls code that throws runs its body through gbn::guard: whatever
// the body throws ends here as a status code and a text for gbn_last_error(); nothing unwinds through an extern "C"
// frame into ctypes, cgo or the toolkit's C core.
#pragma once
#include <exception>
#include <new>
#include <stdexcept>
#include <string>
#include <utility>
#include "../../include/gblastn_amd.h"
#include "../../include/gblastn_amd_host.hpp"     // gbn::CBlastException: an exception that carries a status of this library

namespace gbn {
void set_error(const std::string &msg);

// the status an exception becomes: out of memory -> GBN_ERR_NOMEM; gbn::CBlastException -> the status it carries (its
// text already holds gbn_last_error() of the failing call); everything else -> GBN_ERR_INTERNAL
template <class R, class F> inline R guard_as(const char *fn, R on_nomem, R on_other, F &&body) noexcept
{
    try { return body(); }
    catch (const std::bad_alloc &) { try { set_error(std::string(fn) + ": out of memory"); } catch (...) {} return on_nomem; }
    catch (const std::length_error &) { try { set_error(std::string(fn) + ": out of memory (size beyond the container's limit)"); } catch (...) {} return on_nomem; }
    catch (const CBlastException &e) { try { set_error(std::string(fn) + ": " + e.what()); } catch (...) {} return e.code != GBN_OK ? (R)e.code : on_other; }
    catch (const std::exception &e) { try { set_error(std::string(fn) + ": " + e.what()); } catch (...) {} return on_other; }
    catch (...) { try { set_error(std::string(fn) + ": unknown exception"); } catch (...) {} return on_other; }
}
// status-returning entry points
template <class F> inline int guard(const char *fn, F &&body) noexcept
{
    return guard_as<int>(fn, GBN_ERR_NOMEM, GBN_ERR_INTERNAL, std::forward<F>(body));
}
}  // namespace gbn
