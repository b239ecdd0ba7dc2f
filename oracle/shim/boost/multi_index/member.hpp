// TEST INFRASTRUCTURE ONLY (oracle/shim): see boost/multi_index_container.hpp
#include <boost/multi_index_container.hpp>
