// TEST INFRASTRUCTURE ONLY (oracle/shim): the sliver of Boost.PropertyMap that the reference's
// Common/ContigNode.h and Graph/*.h name when AdjList is compiled as the parity oracle of the
// k-1 overlap stage (oracle/Makefile `ref`).  Tags and the one helper; not a port of anything.
#ifndef ABG_SHIM_BOOST_PROPERTY_MAP_HPP
#define ABG_SHIM_BOOST_PROPERTY_MAP_HPP
#include <boost/graph/properties.hpp>
namespace boost {
template <class Reference, class Map> struct put_get_helper {};
template <class Map, class Reference, class K>
inline Reference get(const put_get_helper<Reference, Map>& m, const K& k) { return static_cast<const Map&>(m)[k]; }
template <class Map, class Reference, class K, class V>
inline void put(const put_get_helper<Reference, Map>& m, const K& k, const V& v) { static_cast<const Map&>(m)[k] = v; }
}
#endif
