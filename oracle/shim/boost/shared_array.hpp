/* Minimal Boost-subset glue so the unmodified reference compiles without Boost.
 * TEST INFRASTRUCTURE ONLY (oracle/_ref build). */
#ifndef ORACLE_SHIM_BOOST_SHARED_ARRAY_HPP
#define ORACLE_SHIM_BOOST_SHARED_ARRAY_HPP
#include <memory>
namespace boost {
template <typename T>
class shared_array {
	std::shared_ptr<T> p_;
public:
	shared_array() {}
	explicit shared_array(T* p) : p_(p, std::default_delete<T[]>()) {}
	T* get() const { return p_.get(); }
	T& operator[](std::ptrdiff_t i) const { return p_.get()[i]; }
};
}
#endif
