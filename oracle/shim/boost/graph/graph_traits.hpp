/* Minimal Boost.Graph-subset glue so the unmodified reference compiles without Boost.
 * TEST INFRASTRUCTURE ONLY (oracle/_ref build). */
#ifndef ORACLE_SHIM_BOOST_GRAPH_TRAITS_HPP
#define ORACLE_SHIM_BOOST_GRAPH_TRAITS_HPP
#include <utility>
#include <tuple>
#include <deque>
namespace boost {

/* Boost's primary template takes the associated types from the graph class itself; the reference's
 * DirectedGraph / ContigGraph (AdjList oracle) rely on that, RollingBloomDBG.h specialises it. */
template <typename G> struct graph_traits {
	typedef typename G::vertex_descriptor vertex_descriptor;
	typedef typename G::edge_descriptor edge_descriptor;
	typedef typename G::adjacency_iterator adjacency_iterator;
	typedef typename G::out_edge_iterator out_edge_iterator;
	typedef typename G::in_edge_iterator in_edge_iterator;
	typedef typename G::vertex_iterator vertex_iterator;
	typedef typename G::edge_iterator edge_iterator;
	typedef typename G::directed_category directed_category;
	typedef typename G::edge_parallel_category edge_parallel_category;
	typedef typename G::traversal_category traversal_category;
	typedef typename G::vertices_size_type vertices_size_type;
	typedef typename G::edges_size_type edges_size_type;
	typedef typename G::degree_size_type degree_size_type;
	static vertex_descriptor null_vertex() { return G::null_vertex(); }
};

struct directed_tag {};
struct undirected_tag {};
struct bidirectional_tag : directed_tag {};
struct allow_parallel_edge_tag {};
struct disallow_parallel_edge_tag {};
struct incidence_graph_tag {};
struct adjacency_graph_tag {};
struct bidirectional_graph_tag : incidence_graph_tag {};
struct vertex_list_graph_tag {};
struct edge_list_graph_tag {};

/* source/target for graphs whose edge_descriptor is a std::pair of vertices */
template <typename V, typename G>
inline V source(const std::pair<V, V>& e, const G&) { return e.first; }
template <typename V, typename G>
inline V target(const std::pair<V, V>& e, const G&) { return e.second; }

/* boost::tie for the (begin,end) iterator-pair idiom and (value,code) pairs */
template <typename A, typename B>
inline std::tuple<A&, B&> tie(A& a, B& b) { return std::tuple<A&, B&>(a, b); }

namespace tuples {
struct swallow_assign {
	template <typename T> const swallow_assign& operator=(const T&) const { return *this; }
};
static const swallow_assign ignore = swallow_assign();
}

template <typename T>
class queue {
	std::deque<T> q_;
public:
	void push(const T& t) { q_.push_back(t); }
	void pop() { q_.pop_front(); }
	T& top() { return q_.front(); }
	const T& top() const { return q_.front(); }
	T& front() { return q_.front(); }
	bool empty() const { return q_.empty(); }
	size_t size() const { return q_.size(); }
};

template <typename T> inline void function_requires() {}
namespace detail {
inline bool is_directed(directed_tag) { return true; }
inline bool is_directed(undirected_tag) { return false; }
}

} // namespace boost

/* The reference calls source()/target() unqualified (Graph/ExtendPath.h,
 * Graph/BreadthFirstSearch.h); with real Boost, ADL finds them through the
 * graph tags. Export them to the global namespace instead. */
using boost::source;
using boost::target;
#endif
