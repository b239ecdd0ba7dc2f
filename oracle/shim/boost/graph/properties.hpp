/* Minimal Boost.Graph property glue. TEST INFRASTRUCTURE ONLY (oracle/_ref build). */
#ifndef ORACLE_SHIM_BOOST_GRAPH_PROPERTIES_HPP
#define ORACLE_SHIM_BOOST_GRAPH_PROPERTIES_HPP
#include <boost/graph/graph_traits.hpp>
namespace boost {
struct no_property {};
enum default_color_type { white_color, gray_color, green_color, red_color, black_color };
template <typename C> struct color_traits;
template <> struct color_traits<default_color_type> {
	static default_color_type white() { return white_color; }
	static default_color_type gray() { return gray_color; }
	static default_color_type green() { return green_color; }
	static default_color_type red() { return red_color; }
	static default_color_type black() { return black_color; }
};
template <typename PM> struct property_traits;
#ifndef ABG_SHIM_PMAP_TAGS
#define ABG_SHIM_PMAP_TAGS
struct readable_property_map_tag {};
struct writable_property_map_tag {};
struct read_write_property_map_tag : readable_property_map_tag, writable_property_map_tag {};
struct lvalue_property_map_tag : read_write_property_map_tag {};
#endif
template <typename G, typename Tag, typename Enable = void> struct property_map;

enum edge_bundle_t { edge_bundle };
enum edge_name_t { edge_name };
enum edge_weight_t { edge_weight };
enum vertex_bundle_t { vertex_bundle };
enum vertex_index_t { vertex_index };
enum vertex_name_t { vertex_name };
/* Boost takes these from the graph class (G::vertex_property_type, ...); a graph that names none gets no_property. */
namespace shim_detail {
template <typename T> struct voider { typedef void type; };
#define ABG_SHIM_NESTED_OR_NONE(TRAIT, NESTED) \
	template <typename G, typename = void> struct TRAIT##_impl { typedef no_property type; }; \
	template <typename G> struct TRAIT##_impl<G, typename voider<typename G::NESTED>::type> { typedef typename G::NESTED type; };
ABG_SHIM_NESTED_OR_NONE(edge_bundle_type, edge_bundled)
ABG_SHIM_NESTED_OR_NONE(vertex_bundle_type, vertex_bundled)
ABG_SHIM_NESTED_OR_NONE(edge_property, edge_property_type)
ABG_SHIM_NESTED_OR_NONE(vertex_property, vertex_property_type)
#undef ABG_SHIM_NESTED_OR_NONE
}
template <typename G> struct edge_bundle_type : shim_detail::edge_bundle_type_impl<G> {};
template <typename G> struct vertex_bundle_type : shim_detail::vertex_bundle_type_impl<G> {};
template <typename G> struct edge_property : shim_detail::edge_property_impl<G> {};
template <typename G> struct vertex_property : shim_detail::vertex_property_impl<G> {};
}
#define BOOST_INSTALL_PROPERTY(KIND, NAME) typedef int oracle_shim_##KIND##_##NAME##_installed
#endif
