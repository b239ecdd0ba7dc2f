// TEST INFRASTRUCTURE ONLY (oracle/shim): what Common/InsOrderedMap.h (pulled in by DataBase/DB.h,
// the optional sqlite statistics AdjList.cpp includes unconditionally) asks of Boost.MultiIndex --
// an insertion-ordered sequence.  The ordered index is never looked at by the oracle build (no --db).
#ifndef ABG_SHIM_BOOST_MULTI_INDEX_HPP
#define ABG_SHIM_BOOST_MULTI_INDEX_HPP
#include <vector>
#include <cstddef>
namespace boost { namespace multi_index {
template <class... I> struct indexed_by {};
template <class T = void> struct random_access {};
template <class K> struct ordered_unique {};
template <class C, class M, M C::*P> struct member {};
template <class T, class Indices>
class multi_index_container {
  public:
	typedef typename std::vector<T>::const_iterator iterator;
	typedef iterator const_iterator;
	template <int N> struct nth_index { typedef multi_index_container type; };
	template <int N> const multi_index_container& get() const { return *this; }
	void push_back(const T& x) { v_.push_back(x); }
	iterator begin() const { return v_.begin(); }
	iterator end() const { return v_.end(); }
	std::size_t size() const { return v_.size(); }
	bool empty() const { return v_.empty(); }
	void clear() { v_.clear(); }
	void erase(iterator it) { v_.erase(v_.begin() + (it - v_.begin())); }
  private:
	std::vector<T> v_;
};
} }
#endif
