"""MI355X-native coupling-flow log_prob path."""
