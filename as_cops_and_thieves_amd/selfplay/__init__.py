"""Self-play support around the env core (SURVEY.md 8f "next" rows): policy archive with PFSP
opponent sampling, MAPPO trainer."""
