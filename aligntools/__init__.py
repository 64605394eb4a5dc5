"""aligntools -- MI355X-native drop-in for the DP hot path of r3fang/alignTools."""
