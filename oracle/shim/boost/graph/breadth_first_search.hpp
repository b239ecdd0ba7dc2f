#ifndef ORACLE_SHIM_BOOST_GRAPH_BFS_HPP
#define ORACLE_SHIM_BOOST_GRAPH_BFS_HPP
#include <boost/graph/graph_traits.hpp>
#include <boost/graph/properties.hpp>
#endif
