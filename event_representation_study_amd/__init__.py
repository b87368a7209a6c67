"""MI355X-native event-representation engine (hot path of uzh-rpg/event_representation_study)."""
__version__ = "0.1.0"
