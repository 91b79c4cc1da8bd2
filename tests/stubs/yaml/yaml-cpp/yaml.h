// Declarations-only stand-in for <yaml-cpp/yaml.h> -- SYNTAX CHECK ONLY (see ../../petsc/README.md).
// yaml-cpp is not installed in this image.  The reference's linear-solver sources name YAML::Node in ONE signature
// (createLinSolver, include/petibm/linsolver.h:217-218) and read three scalars from it (src/linsolver/linsolver.cpp:67-72):
// this declares that much, defines nothing, links to nothing, and no parity claim rests on it.
#pragma once
#include <string>

namespace YAML {
class Node {
public:
    Node();
    template <class Key> const Node operator[](const Key &key) const;
    template <class T> T as() const;
    template <class T, class S> T as(const S &fallback) const;
};
}  // namespace YAML
