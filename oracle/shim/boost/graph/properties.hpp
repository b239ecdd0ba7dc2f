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
struct read_write_property_map_tag {};
struct readable_property_map_tag {};

enum edge_bundle_t { edge_bundle };
enum edge_name_t { edge_name };
enum edge_weight_t { edge_weight };
enum vertex_bundle_t { vertex_bundle };
enum vertex_index_t { vertex_index };
enum vertex_name_t { vertex_name };
template <typename G> struct edge_bundle_type { typedef no_property type; };
template <typename G> struct vertex_bundle_type { typedef no_property type; };
template <typename G> struct edge_property { typedef no_property type; };
template <typename G> struct vertex_property { typedef no_property type; };
}
#define BOOST_INSTALL_PROPERTY(KIND, NAME) typedef int oracle_shim_##KIND##_##NAME##_installed
#endif
