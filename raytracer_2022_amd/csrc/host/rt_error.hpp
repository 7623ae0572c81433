// rt_error.hpp — thread-local error string behind rt_last_error() (include/rt2022.h).
#ifndef RT2022_RT_ERROR_HPP
#define RT2022_RT_ERROR_HPP
#include <string>
namespace rt2022 {
void set_error(const std::string &m);
}
#endif
