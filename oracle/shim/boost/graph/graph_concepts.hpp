#ifndef ORACLE_SHIM_BOOST_GRAPH_CONCEPTS_HPP
#define ORACLE_SHIM_BOOST_GRAPH_CONCEPTS_HPP
#include <boost/graph/graph_traits.hpp>
#endif
